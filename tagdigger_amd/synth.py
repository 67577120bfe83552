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
    # config 5 as SURVEY App. B writes it: degenerate cut site (two concrete sites, tags carry theirs: the
    # multi-cut-site branch of reference tagdigger_fun.py:227-231), barcodes of 4-10 bp, 5 % of the markers
    # tri-allelic -- written as Merged-format rows and expanded by readTags_Merged (:592-598) -- and the
    # common cutter's adapter read through in 20 % of the tag-bearing reads (:27-28) for the splitter branch
    5: dict(nreads=1_000_000_000, nbar=384, nmarkers=50_000, seed=5, cutsite="CWGC", bclen=(4, 10),
            triallelic_pct=5, adapter_pct=20),
}

# what a read runs into behind a short fragment: the rest of the common cutter's site + its adapter
# (adapters['PstI-MspI-Hall'][0] of the reference, tagdigger_fun.py:27-28: 'CCG^G' + top strand)
READ_THROUGH = "CCG" + "CTCAGGCATCACTCGATTCCTCCGTCGTATGCCGTCTTCTGCTTG"


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


def merged_rows(tags, nmarkers, rng, triallelic_pct):
    """The biallelic tag pairs as Merged-format rows (marker name, sequence with the variable site as
    [A/C]); `triallelic_pct` % of the markers get a third allele at the same site ([A/C/T])."""
    rows = []
    third = rng.integers(0, 100, nmarkers) < triallelic_pct
    pick = rng.integers(0, 2, nmarkers)
    for m in range(nmarkers):
        a, b = tags[2 * m], tags[2 * m + 1]
        pos = next(i for i in range(len(a)) if a[i] != b[i])
        alleles = [a[pos], b[pos]]
        if third[m]:
            alleles.append([x for x in "ACGT" if x not in alleles][int(pick[m])])
        rows.append(("M%d" % m, a[:pos] + "[" + "/".join(alleles) + "]" + a[pos + 1:]))
    return rows


def zipf_cdf(n, s, rng):
    """n ascending 64-bit thresholds (the last one 2^64-1) of a Zipf(s) law whose ranks are dealt to the
    indices by a seeded permutation (hot tags are scattered over the columns, as in real data)."""
    w = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** s
    w = w[rng.permutation(n)]
    c = np.cumsum(w / w.sum())
    t = np.minimum(np.floor(c * 2.0 ** 64), 2.0 ** 64 - 2049).astype(np.uint64)   # (float64 cannot reach 2^64-1)
    t = np.maximum.accumulate(t)
    t[-1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return np.ascontiguousarray(t)


class SynthConfig:
    def __init__(self, nreads, nbar, nmarkers, seed, cutsite="TGCAG", bclen=(4, 8), read_len=100, body=59,
                 skew=0.0, triallelic_pct=0, adapter_pct=0):
        self.nreads, self.seed, self.cutsite, self.read_len = nreads, seed, cutsite, read_len
        self.cutsites = enumerate_cut_sites(cutsite)
        rng = np.random.default_rng(seed)
        self.barcodes = make_barcodes(rng, nbar, bclen[0], bclen[1], self.cutsites)
        self.tags = make_tags(rng, nmarkers, self.cutsites, body)
        self.tag_names = None
        if triallelic_pct:
            # through the reader the reference's CLI would use: a Merged-format file, expanded on the host
            import csv
            import os
            import tempfile
            from . import tagdigger_fun as tf
            rows = merged_rows(self.tags, nmarkers, rng, triallelic_pct)
            fd, path = tempfile.mkstemp(suffix=".csv")
            try:
                with os.fdopen(fd, "w", newline="") as fh:
                    w = csv.writer(fh)
                    w.writerow(["Marker name", "Tag sequence"])
                    w.writerows(rows)
                self.tag_names, self.tags = tf.readTags_Merged(path)
            finally:
                os.unlink(path)
        self.skew = float(skew)
        self.adapter_pct = int(adapter_pct)
        # draw tables of the skewed variant (Zipf(s) over the tags, Zipf(s/2) over the barcodes); kept alive here
        self.tag_cdf = zipf_cdf(len(self.tags), self.skew, rng) if self.skew else None
        self.bar_cdf = zipf_cdf(len(self.barcodes), self.skew / 2, rng) if self.skew else None
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

    def params(self, cls=None):
        """td_synth_params (include/td_synth_spec.h); `cls`: the ctypes mirror to fill (the binding's by default)."""
        ad = READ_THROUGH[:64].encode() if self.adapter_pct else b""
        return (cls or B.SynthParams)(
            self.seed, len(self.barcodes), len(self.tags), len(self.cutsites), self.read_len, len(self.cutsites[0]),
            self.tag_stride, self.adapter_pct, len(ad),
            self.tag_cdf.ctypes.data if self.tag_cdf is not None else None,
            self.bar_cdf.ctypes.data if self.bar_cdf is not None else None, ad)

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
